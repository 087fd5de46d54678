#!/bin/bash
# round 3, GPU call C: new tests (fallback, cross-process reproducibility), optimize-vs-oracle exploration, the new bench line
set -x
O=gpurun_out/r3c; mkdir -p $O
timeout 600 python -m pytest tests/test_edge_gpu.py tests/test_repro_gpu.py tests/test_bench_gpu.py -m gpu -x -q > $O/pytest_new.log 2>&1; echo "pytest_new rc=$?"
tail -15 $O/pytest_new.log
timeout 900 python tools/optimize_parity.py 24 5 18 > $O/optimize_parity_24.jsonl 2> $O/optimize_parity_24.err; cat $O/optimize_parity_24.jsonl
timeout 900 python tools/optimize_parity.py 32 6 18 > $O/optimize_parity_32.jsonl 2> $O/optimize_parity_32.err; cat $O/optimize_parity_32.jsonl
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json
timeout 600 python bench.py --loop iterate --no-extra --no-cpu-baseline > $O/bench_iterate.json 2> $O/bench_iterate.err; cat $O/bench_iterate.json
