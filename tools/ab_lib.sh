#!/bin/bash
# A/B of two BUILDS of the engine on the headline workload, interleaved (boxes drift): tools/ab_lib.sh tag libA.so libB.so [rounds]
tag=$1; A=$2; B=$3; n=${4:-3}
out=gpurun_out/ab_${tag}.txt; mkdir -p gpurun_out; : > $out
for r in $(seq 1 $n); do for lib in $A $B; do
  PSGSDF_ENGINE_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-extra --reps 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
rf = d['roofline']
print('$lib', round(d['value'], 1), [round(x, 1) for x in d['spread']['values']], 'solve avg us', round(1e3 * rf['avg_launch_ms'], 1), 'per pass', round(rf.get('us_per_pass') or 0, 2), 'fixed', round(rf.get('fixed_us') or 0, 1))
" >> $out
done; done
cat $out
