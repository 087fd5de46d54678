#!/bin/bash
# A/B of environment settings on the headline workload: tools/ab_env.sh tag "A=1 B=2" "A=3" ...   (one bench line per setting -> gpurun_out/ab_<tag>.txt)
tag=$1; shift
out=gpurun_out/ab_${tag}.txt; mkdir -p gpurun_out; : > $out
for v in "$@"; do
  env $v python bench.py --no-cpu-baseline --no-extra --reps 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
k = d.get('kernels_sync_pass') or {}
print('$v', round(d['value'], 1), [round(x, 1) for x in d['spread']['values']], {n: round(1e3 * t, 1) for n, t in k.items()} if isinstance(k, dict) else k)
" >> $out
done
cat $out
