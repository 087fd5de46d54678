#!/bin/bash
set -x
O=gpurun_out/r3f; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_slab_gpu.py tests/test_bench_gpu.py -m gpu -x -q > $O/pytest_new.log 2>&1; echo "pytest_new rc=$?"; tail -25 $O/pytest_new.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['iterate_ms_per_step'], d['config']['loop'][:40], d['sync_stats'])"
timeout 1200 python bench.py --strong --steps 5 --warmup 2 --no-breakdown > $O/bench_strong1.json 2> $O/bench_strong1.err; echo "strong rc=$?"; cat $O/bench_strong1.json | cut -c1-1500; tail -3 $O/bench_strong1.err
