#!/bin/bash
set -x
O=gpurun_out/r3g; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_edge_gpu.py tests/test_knobs_gpu.py tests/test_repro_gpu.py tests/test_host_mirror_gpu.py -m gpu -x -q > $O/pytest_new.log 2>&1; echo "pytest_new rc=$?"; tail -25 $O/pytest_new.log
for sp in 1 0; do PSGSDF_SPECULATE=$sp timeout 900 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $O/bench_spec$sp.json 2> $O/bench_spec$sp.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$O/bench_spec$sp.json')); print('spec=$sp', d['value'], d['ms_per_step'], d['iterate_ms_per_step'], d['config']['loop'][:60], d['sync_stats'])"; done
timeout 900 python tools/soak.py --models LED,SH1,SH2 --n 64 --frames 8 --procs 4 --reps 20 --variants default,spec0,fold0,persist0 --out $O/soak > $O/soak.log 2>&1; echo "soak rc=$?"; tail -c 1500 $O/soak.log
