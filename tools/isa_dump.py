#!/usr/bin/env python3
"""isa_dump.py FILE.s KERNEL LO HI | isa_dump.py FILE.s KERNEL find REGEX - the instructions
LO..HI of one kernel in `hipcc -S -gline-tables-only --cuda-device-only` output, numbered as
tools/isa_loops.py numbers them, each with the source line it comes from; or the instructions
that match REGEX.  A tuning aid, not part of the product."""
import re,sys
path,kern=sys.argv[1],sys.argv[2]
lines=open(path).read().split('\n')
start=next(i for i,l in enumerate(lines) if l.startswith(kern+':'))
end=next(i for i in range(start,len(lines)) if 's_endpgm' in lines[i])
idx=0;cur=0;rows=[]
for l in lines[start:end+1]:
    t=l.strip()
    m=re.match(r"\.loc\s+\d+\s+(\d+)",t)
    if m: cur=int(m.group(1)); continue
    if re.match(r"^\.LBB\d+_\d+:",l):
        rows.append((idx,cur,l.split(';')[0])); continue
    if not t or t.startswith(';') or t.startswith('.'): continue
    rows.append((idx,cur,t)); idx+=1
if sys.argv[3]=='find':
    for n,c,t in rows:
        if re.search(sys.argv[4],t): print(n,c,t)
else:
    lo,hi=int(sys.argv[3]),int(sys.argv[4])
    for n,c,t in rows:
        if lo<=n<=hi: print(n,c,t)
