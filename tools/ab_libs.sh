#!/bin/bash
# A/B/.. of several BUILDS of the engine on the headline workload, interleaved (boxes drift): tools/ab_libs.sh tag rounds libA.so libB.so [libC.so ...]
# one line per run: it/s (median of 3 repetitions), the spread, and the kernels' us per iteration from the synchronised pass
tag=$1; n=$2; shift 2
out=gpurun_out/ab_${tag}.txt; mkdir -p gpurun_out; : > $out
for r in $(seq 1 $n); do for lib in "$@"; do
  PSGSDF_ENGINE_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-extra --reps 3 $AB_BENCH_ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
k = d.get('kernels_sync_pass') or {}
print('$lib', round(d['value'], 1), [round(x, 1) for x in d['spread']['values']], {n: round(1e3 * t, 1) for n, t in k.items() if t > 0.003} if isinstance(k, dict) else k)
" >> $out
done; done
cat $out
