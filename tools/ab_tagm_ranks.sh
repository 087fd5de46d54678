for t in 1 2 1 2; do
for cfg in "2 --grid 256 --frames 50" "8 --grid 96 --frames 50"; do
set -- $cfg; n=$1; shift
PSGSDF_PCG_TAGM=$t PSGSDF_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus $n "$@" --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); m=d.get('multi_gpu',{})
print('TAGM=$t ranks=$n', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'degraded', d.get('degraded'), 'xsolves', m.get('cross_rank_solves'), 'fallbacks', m.get('persist_fallbacks'), 'self_check', d.get('self_check'))
"
done; done
