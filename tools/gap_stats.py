#!/usr/bin/env python3
"""GPU idle-gap analysis of a rocprofv3 rocpd database: which kernel->kernel transitions leave the GPU idle."""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
ev = [(s, e, re.sub(r'.*psg::(k_\w+).*', r'\1', n)) for n, s, e in cur.execute("select name,start,end from kernels")]
ev.sort()
ev2 = ev[int(len(ev) * 0.5):]
gap = collections.Counter(); cnt = collections.Counter(); busy = 0; samples = collections.defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(ev2, ev2[1:]):
    g = max(0, s1 - e0); samples[(n0, n1)].append(g); gap[(n0, n1)] += g; cnt[(n0, n1)] += 1; busy += e0 - s0
tot = ev2[-1][1] - ev2[0][0]
print(f"window {tot/1e3:.0f} us, busy {busy/1e3:.0f} us ({100*busy/tot:.0f} %), idle {sum(gap.values())/1e3:.0f} us")
for k, v in gap.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    q = sorted(samples[k])
    print(f"  {k[0]:>32} -> {k[1]:<32} {v/1e3:8.1f} us over {cnt[k]:4d}  avg {v/1e3/cnt[k]:6.2f}  median {q[len(q)//2]/1e3:5.2f}  p90 {q[int(len(q)*0.9)]/1e3:5.2f}  max {q[-1]/1e3:6.2f}")
