#!/usr/bin/env python3
"""isa_live.py FILE.s KERNEL [inst_index...] - VGPR liveness over the ISA text of one
kernel (`hipcc -S --cuda-device-only`): defs and uses per instruction, a backward data-flow
pass over the branch graph (writes kill; EXEC-partial writes count as full writes, so the
figures inside divergent code are a lower bound).  Prints the live VGPRs at each requested
instruction index (the indices of tools/isa_dump.py / tools/isa_loops.py) and the maximum.
This is how round 5 found 55 registers holding hoisted constants through every hot loop of the
compress kernel.  A tuning aid, not part of the product."""
import re, sys, pickle
path, kern = sys.argv[1], sys.argv[2]
want = [int(x) for x in sys.argv[3:]]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(kern + ":"))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
insts = []; labels = {}
for l in lines[start:end+1]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m: labels[m.group(1)] = len(insts); continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."): continue
    t = t.split(";")[0].strip()
    insts.append(t)
def regs(tok):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1): out += list(range(int(m.group(1)), int(m.group(2))+1))
        else: out.append(int(m.group(3)))
    return out
N = len(insts)
defs = [set() for _ in range(N)]; uses = [set() for _ in range(N)]; succ = [[] for _ in range(N)]
for i, t in enumerate(insts):
    op = t.split()[0]
    ops = t[len(op):].split(",") if len(t) > len(op) else []
    ops = [o.strip() for o in ops]
    m = re.match(r"s_cbranch\w*\s+(\.LBB\d+_\d+)", t); b = re.match(r"s_branch\s+(\.LBB\d+_\d+)", t)
    if b: succ[i] = [labels[b.group(1)]]
    elif m: succ[i] = [labels[m.group(1)]] + ([i+1] if i+1 < N else [])
    elif op == "s_endpgm": succ[i] = []
    else: succ[i] = [i+1] if i+1 < N else []
    if not ops: continue
    # which operand(s) are destinations
    nd = 0
    if op.startswith("v_") and not op.startswith(("v_cmp", "v_cmpx")): nd = 1
    if op.startswith(("v_readlane", "v_readfirstlane")): nd = 0
    if op.startswith(("ds_read", "global_load", "scratch_load", "buffer_load", "flat_load")) or "_rtn" in op or op.startswith("ds_bpermute") or op.startswith("ds_permute") or op.startswith("ds_swizzle"): nd = 1
    if op.startswith("v_writelane"):  # read-modify-write of dst
        d = regs(ops[0]); defs[i] = set(); uses[i] = set(d)
        for o in ops[1:]: uses[i] |= set(regs(o))
        continue
    if op.startswith(("v_mad_u64_u32", "v_mad_i64_i32")): nd = 1  # second dst is sgpr/vcc
    for k, o in enumerate(ops):
        r = set(regs(o))
        if k < nd: defs[i] |= r
        else: uses[i] |= r
    if op.startswith(("v_cmp",)) and ops and regs(ops[0]) and "sdwa" not in t and op.endswith("_e32") is False:
        pass
    # dpp/sdwa with partial dst preserve: treat dst as also used when 'dst_unused:UNUSED_PRESERVE' or dpp bound_ctrl absent (conservative)
    if "dpp" in t or "UNUSED_PRESERVE" in t or op.startswith(("v_fmac", "v_mac", "v_dot", "v_mfma")):
        uses[i] |= defs[i]
livein = [set() for _ in range(N)]
changed = True
while changed:
    changed = False
    for i in range(N-1, -1, -1):
        out = set()
        for s in succ[i]: out |= livein[s]
        ni = uses[i] | (out - defs[i])
        if ni != livein[i]: livein[i] = ni; changed = True
mx = max(range(N), key=lambda i: len(livein[i]))
print("max live", len(livein[mx]), "at", mx, insts[mx])
for w in want:
    print(w, len(livein[w]), insts[w], sorted(livein[w]))

