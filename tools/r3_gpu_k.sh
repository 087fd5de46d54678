#!/bin/bash
set -x
O=gpurun_out/r3k; mkdir -p $O
export PSGSDF_COMMIT=c1937a0
for i in 1 2 3 4 5; do timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_$i.log 2>&1; echo "pytest $i rc=$?"; tail -1 $O/pytest_$i.log; done
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 1500 python tools/soak.py --models LED,SH1,SH2 --n 64 --frames 8 --procs 10 --reps 40 --variants default,spec0,fold0,poll0,persist0 --out $O/soak > $O/soak.log 2>&1; echo "soak rc=$?"
timeout 1500 python tools/soak.py --models SH1,LED,SH2 --n 256 --frames 50 --width 640 --height 480 --procs 3 --reps 3 --variants default,spec0,persist0 --timeout 1400 --out $O/soak_big > $O/soak_big.log 2>&1; echo "soak_big rc=$?"
for i in 1 2 3; do timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_$i.json 2> $O/bench_$i.err; python -c "
import json; d=json.load(open('$O/bench_$i.json')); print(d['value'], d['ms_per_step'], d['iterate_ms_per_step'], d['extra']['LED']['value'], d['extra']['SH2']['value'], d['roofline']['frac'])"; done
