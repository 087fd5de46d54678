#!/bin/bash
set -x
O=gpurun_out/r3d; mkdir -p $O
for d in 10 30 100; do timeout 900 python tools/optimize_parity.py 24 5 20 $d > $O/optimize_parity_24_d$d.jsonl 2> $O/optimize_parity_d$d.err; cat $O/optimize_parity_24_d$d.jsonl | cut -c1-900; done
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json; tail -5 $O/bench.err
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
